"""protopformer_amd: MI355X-native (gfx950) implementation of ProtoPFormer's training hot path.

Python here is host-side plumbing (tensor allocation, streams, autograd wiring, the reference's
PPNet API); all arithmetic of the path runs in hand-written HIP kernels reached through the C ABI
declared in include/ppf_hip.h (protopformer_amd/lib/libppf_hip.so)."""
__version__ = "0.1.0"
