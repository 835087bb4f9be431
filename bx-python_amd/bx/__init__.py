"""
Drop-in overlay for bx-python's two hot-path extension modules.

Put this directory FIRST on ``sys.path`` (``PYTHONPATH=.../bx-python_amd``): then
``bx.bitset`` and ``bx.intervals.intersection`` resolve to the MI355X-backed
modules in this package, while every other ``bx.*`` module (bitset_builders,
intervals.io, cookbook, ...) still resolves to an installed bx-python, if there
is one, because the package path is extended over all ``bx`` directories found
later on ``sys.path`` (reference layout: lib/bx/__init__.py).
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
__version__ = "0.14.0+bxmi"
