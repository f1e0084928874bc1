"""dostransformer_amd — MI355X-native DOSTransformer hot path.

Importing the package never touches the GPU or the HIP library; the first op call
loads ``csrc/libdosx.so`` and raises loudly if it is missing.
"""
__version__ = "0.1.0"


def install_dropin() -> None:
    """Register this package's modules under the reference's import paths (``layers``,
    ``embedder_phDOS``, ``embedder_eDOS``) so that the reference drivers' own import lines
    (`main_phDOS.py:67`, `main_eDOS.py:68`, `DOSTransformer.py:6`) resolve to the MI355X build."""
    import importlib
    import sys
    for name in ("layers", "layers.transformer", "layers.multihead_attention",
                 "embedder_phDOS", "embedder_phDOS.DOSTransformer_phonon", "embedder_phDOS.graphnetwork_phonon",
                 "embedder_eDOS", "embedder_eDOS.DOSTransformer", "embedder_eDOS.graphnetwork"):
        sys.modules[name] = importlib.import_module("dostransformer_amd." + name)
