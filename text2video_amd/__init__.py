"""text2video_amd -- MI355X-native frame synthesis for Text2Video (vid2vid pose->RGB generator).

Everything on the device goes through the C ABI of lib/libt2v_hip.so (include/t2v.h); there is
no CPU fallback.  See DESIGN.md.
"""
__version__ = "0.1.0"
