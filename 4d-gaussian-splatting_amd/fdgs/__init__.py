"""MI355X-native hot path of 4D Gaussian splatting: HexPlane deformation field +
tile rasterizer behind the reference's Python surface.  Compute lives in
libfdgs.so (gfx950 HIP kernels, C ABI in include/fdgs.h)."""
