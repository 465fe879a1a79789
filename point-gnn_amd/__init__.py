"""pointgnn_amd -- MI355X-native (gfx950) implementation of the Point-GNN hot
path: fixed-radius graph construction + scatter-max message passing, behind the
reference's own operator surface.  See DESIGN.md / INTEGRATION.md.

Sub-modules (imported lazily; `graph_gen`, `gnn`, `models` need torch + the HIP
library, the others are pure Python):
  synthetic  configs  weights  tf_bundle  build  graph_gen  gnn  models
"""
__version__ = "0.1.0"
