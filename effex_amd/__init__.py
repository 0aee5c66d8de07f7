"""effex_amd — MI355X-native F/X hot path behind effex's Correlator call surface.

Importing the package does not load the HIP library; ``effex_amd._lib.load()`` does, and it raises
if ``effex_amd/csrc/libfxcorr.so`` is missing — there is no CPU fallback.
"""
__all__ = ["Correlator", "FxPlan", "design_window"]


def __getattr__(name):
    if name == "Correlator":
        from .correlator import Correlator
        return Correlator
    if name == "FxPlan":
        from .plan import FxPlan
        return FxPlan
    if name == "design_window":
        from .window import design_window
        return design_window
    raise AttributeError(name)
