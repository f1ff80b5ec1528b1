"""dgpmp2_amd -- MI355X-native inner Gauss-Newton solver behind the dGPMP2 planner API.

Hot path: hand-written HIP kernels (dgpmp2_amd/csrc) behind the C-ABI of include/dgpmp2_hip.h.
Host side: Python mirrors of the reference's `diff_gpmp2.gpmp2` planner interface (same class names,
constructor dicts, method signatures and return tuples).  There is no CPU fallback: importing the
planner classes without the built HIP library raises ImportError.
"""
__version__ = '0.1.0'
