"""meshflow_amd: MI355X-native implementation of MeshFlowStabilizer's dense inner path
(Jacobi temporal smoothing + per-cell mesh warp).  See DESIGN.md."""
from .stabilizer import DegenerateMeshError, MeshFlowStabilizer  # noqa: F401

__all__ = ['MeshFlowStabilizer', 'DegenerateMeshError']
