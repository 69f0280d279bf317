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


def render_sequence(views, pc, deforms, bg_color, keep_renders=False, streams=1, **render_kwargs):
    """The loop of render.py:74-79 over `views`: returns (render_images, render_list) - the uint8 [H,W,3] frames and,
    with keep_renders, the float renders the reference also keeps for writing the PNGs.

    Frames are independent, and each one makes the host wait once (the length of its tile lists comes back from the
    device before the sort can be enqueued).  With `streams` > 1 consecutive frames alternate between that many
    streams, so that the GPU has another frame's work queued during the round trip; measured at 1352 x 1014 with
    300k Gaussians this is within the run-to-run spread of one stream (1450-1520 frames per second either way: the
    round trip is about 0.05 of a frame's 0.65 ms), so one stream is the default."""
    from .renderer import render
    views = list(views)
    if not views:
        return [], []
    sink = None
    render_list = []
    dev = None
    lanes = None
    with torch.no_grad():
        for i, view in enumerate(views):
            if lanes is None:
                dev = torch.device(bg_color.device if isinstance(bg_color, torch.Tensor) else "cuda:0")
                caller = torch.cuda.current_stream(dev)
                lanes = [torch.cuda.Stream(dev) for _ in range(max(1, int(streams)))] if streams > 1 else [caller]
                for s in lanes:
                    s.wait_stream(caller)           # the parameters may still be in flight on the caller's stream
            with torch.cuda.stream(lanes[i % len(lanes)]):
                rendering = render(view, pc, deforms, bg_color, **render_kwargs)["render"]
                if sink is None:
                    sink = FrameSink(len(views), rendering.shape[1], rendering.shape[2], rendering.shape[0],
                                     device=rendering.device)
                sink.push(rendering)
            if keep_renders:
                render_list.append(rendering)
    images = sink.frames()
    for s in lanes:
        s.synchronize()                             # the renders are the caller's to use on any stream
    return images, render_list
