"""MI355X-native drop-in for the hot path of nmwsharp/diffusion-net.

Mirrors the reference package surface used by the experiment scripts
(`import diffusion_net; diffusion_net.layers.DiffusionNet(...)`,
reference src/diffusion_net/__init__.py:1-3).  Importing the package never needs the GPU or
the HIP library; the first hot-path op does, and raises if either is missing."""
from . import utils, geometry, layers, synthetic, batch, ops, graphs, autograph  # noqa: F401
from .batch import MeshBatch, GatherPattern  # noqa: F401
from .layers import DiffusionNet  # noqa: F401
