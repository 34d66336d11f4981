"""nfst_amd -- MI355X-native lattice engine for nFST (forward-backward, log-Z,
posteriors, Viterbi, posterior sampling, arc-score gathers) behind the reference's
Python call sites.  See DESIGN.md / INTEGRATION.md."""
from . import synth  # noqa: F401  (numpy only)

__all__ = ["synth", "LatticeBatch", "ops"]


def __getattr__(name):
    # the HIP library is loaded on first use so that `import nfst_amd.synth` works anywhere
    if name == "LatticeBatch":
        from .lattice import LatticeBatch
        return LatticeBatch
    if name == "ops":
        import importlib
        return importlib.import_module(".ops", __name__)
    raise AttributeError(name)
