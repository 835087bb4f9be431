"""
Drop-in overlay for bx-python's two hot-path extension modules.

Put this directory FIRST on ``sys.path`` (``PYTHONPATH=.../bx-python_amd``): then
``bx.bitset``, ``bx.intervals.intersection`` / ``cluster`` and their batch-aware
callers (``bx.bitset_builders``, ``bx.intervals.io``, ``bx.tabular.io``,
``bx.intervals.operations.*``) resolve to the MI355X-backed modules in this
package, while every other ``bx.*`` module (cookbook, align, seq, the
``concat`` operation, ...) still resolves to an installed bx-python, if there
is one, because the package paths are extended over all ``bx`` directories
found later on ``sys.path`` (reference layout: lib/bx/__init__.py).
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
__version__ = "0.14.0+bxmi"
