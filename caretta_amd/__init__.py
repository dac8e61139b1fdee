"""caretta_amd -- MI355X-native all-vs-all pairwise structural alignment (the hot path of
TurtleTools/caretta) behind the reference's own Python surface.

Modules mirror the reference's: ``dynamic_time_warping``, ``score_functions``,
``superposition_functions``, ``neighbor_joining``, ``helper``, ``multiple_alignment``.
Compute runs in hand-written gfx950 HIP kernels reached through a C ABI
(``include/caretta_hip.h``, ``caretta_amd/csrc``); there is no CPU fallback.
"""
__version__ = "0.1.0"
