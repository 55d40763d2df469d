"""Where the frame-synthesis path (ops / generator / model) takes device buffers, streams and events from.

PyTorch is plumbing on this path -- an allocator, a stream, an event -- and importing it is the largest single term of the
one-shot command's start-up (0.85 s of 2.8 s on the MI355X box for `vid2vid/test.py` on one utterance; the
reference starts one process per utterance, text2video_audio.sh:37-44).  `vid2vid/test.py` therefore calls use_lean()
before it imports the model: the same modules then run on text2video_amd/leantorch.py, the few torch names they use
implemented over the library's own host-plumbing entry points (include/t2v.h, ABI 14), and torch is never imported.
Everything else (tests, trainer, resident server, multi-GPU runs) gets the real torch.

    from ._xp import torch        # in ops.py / generator.py / model.py
"""
import sys

_resolved = None     # (module, lean?)
_want_lean = False


def use_lean():
    """Ask for the torch-free provider.  Only honoured before the first `from ._xp import torch` and in a process that
    has not imported torch already (then the real one costs nothing more).  Returns whether lean mode is on."""
    global _want_lean
    if _resolved is None and "torch" not in sys.modules:
        _want_lean = True
    return _resolve()[1]


def _resolve():
    global _resolved
    if _resolved is None:
        if _want_lean and "torch" not in sys.modules:
            from . import leantorch
            _resolved = (leantorch, True)
        else:
            import torch
            _resolved = (torch, False)
    return _resolved


def __getattr__(name):
    if name == "torch":
        return _resolve()[0]
    if name == "LEAN":
        return _resolve()[1]
    raise AttributeError(name)
