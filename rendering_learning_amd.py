"""Import alias: the package directory is `rendering-learning_amd/` (a hyphen cannot appear in an
`import` statement), so `import rendering_learning_amd` resolves to it."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("rendering-learning_amd")
