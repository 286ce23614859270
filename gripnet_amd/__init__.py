"""gripnet_amd: MI355X-native supergraph propagation engine for GripNet's forward hot path.

Drop-in for ``from gripnet.layers import homoGraph, interGraph, myGCN, myRGCN`` and
``from gripnet.decoder import multiRelaInnerProductDecoder, multiClassInnerProductDecoder``
(reference: GripNet-pose.py:8,12; GripNet-aminer.py:1; baselines/LP_baselines/rgcn_pose.py:1).
The arithmetic runs in hand-written HIP kernels for gfx950 behind a C ABI
(include/gripnet_hip.h, gripnet_amd/lib/libgripnet_hip.so); importing the package does not
need a GPU, running a layer does.
"""
from .layers import myGCN, myRGCN, homoGraph, interGraph
from .decoder import multiRelaInnerProductDecoder, multiClassInnerProductDecoder
from . import utils, synth, optim

__all__ = ["myGCN", "myRGCN", "homoGraph", "interGraph", "multiRelaInnerProductDecoder",
           "multiClassInnerProductDecoder", "utils", "synth", "optim"]
__version__ = "0.1.0"
