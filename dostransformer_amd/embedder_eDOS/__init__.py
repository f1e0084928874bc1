# (the reference's embedder_eDOS/__init__.py is entirely commented out; classes are imported by module path)
