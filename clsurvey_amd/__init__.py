"""clsurvey_amd — MI355X-native training + importance-weight path of the CLsurvey framework.

Layout: csrc/ (HIP kernels + C ABI, built into libclhip.so), _lib.py (ctypes binding),
ops.py (torch-facing wrappers / autograd bridges), net.py (ParamArena + NetEngine),
optim.py + methods/ (host-side mirror of the reference's optimizers and trainers).
"""
__version__ = "0.1.0"
