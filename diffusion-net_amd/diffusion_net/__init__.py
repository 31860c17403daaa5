"""MI355X-native drop-in for the hot path of nmwsharp/diffusion-net.

Mirrors the reference package surface used by the experiment scripts
(`import diffusion_net; diffusion_net.layers.DiffusionNet(...)`,
reference src/diffusion_net/__init__.py:1-3)."""
from . import synthetic  # noqa: F401
