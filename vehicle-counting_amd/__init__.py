"""MI355X-native detect + track hot path behind kaylode/vehicle-counting's stage interface.

Python side = the reference's own duck-typed operator API (modules/detect.py ImageDetect,
modules/track.py VideoTracker / VideoCounting, networks/deepsort/deep_sort.py DeepSort,
modules/__init__.py CountingPipeline) implemented as ctypes calls into `libvcount_hip.so`
(C ABI declared in include/vcount_hip.h; HIP kernels in csrc/).  Nothing here computes on the CPU
what the reference computes on the accelerator: if the HIP library is missing, construction fails.
"""
__version__ = "0.1.0"
