"""dostransformer_amd — MI355X-native DOSTransformer hot path.

Importing the package never touches the GPU or the HIP library; the first op call
loads ``csrc/libdosx.so`` and raises loudly if it is missing.
"""
__version__ = "0.1.0"
