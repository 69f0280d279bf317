"""8-bit frames of a rendered sequence, as the reference's playback loop collects them
(render.py:74-89: `render_images.append(to8b(rendering).transpose(1, 2, 0))`, to8b of render.py:47 being
`(255 * np.clip(x.cpu().numpy(), 0, 1)).astype(np.uint8)`).

There every frame crosses to the host as 12 bytes per pixel of float32 and is clipped, scaled and truncated by numpy
while the GPU waits.  Here `fdgs_image_to8b` does the conversion on the device (the same fp32 multiply and
truncation: the bytes are identical) and the 3 bytes per pixel travel on a copy stream into a pinned array that holds
the whole sequence, while the next frame is already being rendered.
"""
import torch

from ._lib import lib, ptr, check, require_gpu


class FrameSink:
    """Collects up to `n_frames` rendered [C,H,W] float images as uint8 [H,W,C] numpy arrays.

    push(image) enqueues the conversion on the current stream and the copy on a second one and returns at once;
    frames() waits for the copies and returns the arrays (views of one pinned host tensor)."""

    def __init__(self, n_frames, height, width, channels=3, device="cuda:0"):
        self.shape = (int(channels), int(height), int(width))
        self.device = torch.device(device)
        self.host = torch.empty(int(n_frames), int(height), int(width), int(channels), dtype=torch.uint8,
                                pin_memory=True)
        self.staging = [torch.empty(int(height), int(width), int(channels), dtype=torch.uint8, device=self.device)
                        for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(self.device)
        self.converted = [torch.cuda.Event() for _ in range(2)]
        self.copied = [None, None]
        self.count = 0

    def push(self, image):
        require_gpu(image)
        if tuple(image.shape) != self.shape or image.dtype is not torch.float32:
            raise ValueError(f"expected a float32 image of shape {self.shape}, got {image.dtype} {tuple(image.shape)}")
        if self.count >= self.host.shape[0]:
            raise IndexError("FrameSink is full")
        image = image.detach()
        if not image.is_contiguous():
            image = image.contiguous()
        k = self.count & 1
        cur = torch.cuda.current_stream(self.device)
        if self.copied[k] is not None:
            cur.wait_event(self.copied[k])          # the staging buffer's previous frame has left
        c, h, w = self.shape
        check(lib.fdgs_image_to8b(ptr(image), ptr(self.staging[k]), c, h, w, cur.cuda_stream), "fdgs_image_to8b")
        self.converted[k].record(cur)
        self.copy_stream.wait_event(self.converted[k])
        with torch.cuda.stream(self.copy_stream):
            self.host[self.count].copy_(self.staging[k], non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        self.copied[k] = done
        self.count += 1

    def frames(self):
        self.copy_stream.synchronize()
        return [self.host[i].numpy() for i in range(self.count)]


def render_sequence(views, pc, deforms, bg_color, keep_renders=False, to8b=True, pipelined=True, **render_kwargs):
    """The loop of render.py:74-79 over `views`: returns (render_images, render_list) - the uint8 [H,W,3] frames
    (to8b) and, with keep_renders, the float renders the reference also keeps for writing the PNGs.

    Every frame makes the host wait once: the length of its tile lists comes back from the device before the sort
    can be enqueued, and the GPU has nothing to do from the end of the preprocess kernels until the host has woken
    up and launched the sort (about 0.06 of a frame's 0.63 ms at 1352 x 1014 with 300k Gaussians).  Frames are
    independent, so with `pipelined` the NEXT frame's deformation field and preprocess are enqueued before the host
    waits for the current frame's length: the GPU works on them through the round trip.  The frames are the same
    (same kernels on the same inputs, tests/test_render_gpu.py); two frames' intermediates are alive at a time."""
    from .renderer import prepare, render
    from .rasterizer import rasterize_begin_call, rasterize_finish
    views = list(views)
    if not views:
        return [], []
    sink = None
    render_list = []

    def take(rendering):
        nonlocal sink
        if to8b:
            if sink is None:
                sink = FrameSink(len(views), rendering.shape[1], rendering.shape[2], rendering.shape[0],
                                 device=rendering.device)
            sink.push(rendering)
        if keep_renders:
            render_list.append(rendering)

    with torch.no_grad():
        if not pipelined:
            for view in views:
                take(render(view, pc, deforms, bg_color, **render_kwargs)["render"])
        else:
            def begin(view):
                settings, kwargs, _ = prepare(view, pc, deforms, bg_color, screenspace=False, **render_kwargs)
                return rasterize_begin_call(settings, **kwargs)
            pending = begin(views[0])
            for i in range(len(views)):
                following = begin(views[i + 1]) if i + 1 < len(views) else None
                take(rasterize_finish(pending).color)
                pending = following
    return (sink.frames() if sink is not None else []), render_list
