"""Module name the reference imports (src/gaussian_renderer/__init__.py:14,
src/dataset/dataset_readers.py:185); the implementation is fdgs.rasterizer."""
from fdgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401
