"""dgnn_amd -- MI355X-native (gfx950) engine for the message-passing hot path of raphaelsulzer/dgnn.

Layout: csrc/ (HIP kernels + C ABI -> libdgnn_hip.so), _lib.py (ctypes binding), ops.py (tensor
wrappers), graph.py (plans), functional.py (autograd), learning/ (drop-in model modules with the
reference's names), partition.py (multi-GPU halo exchange), synthetic.py (benchmark graphs).
"""
__version__ = "0.1.0"
