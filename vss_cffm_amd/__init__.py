"""vss_cffm_amd -- MI355X-native (gfx950) implementation of the CFFM decode head's coarse-to-fine
cross-frame attention path (CFFA + CFM, and the CFFM++ prototype attention), behind the
reference's own module / registry names.  See DESIGN.md and include/cffm_hip.h."""
from .modules import (BasicLayer3d3, BasicLayer_cluster, CffmTransformerBlock3d3, Mlp,  # noqa: F401
                      SwinTransformerBlock_cluster, WindowAttention3d3, WindowAttention_cluster)
