from .transformer import TransformerEncoder  # noqa: F401  (same export as the reference's layers/__init__.py:1)
