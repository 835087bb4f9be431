"""bx.tabular: `io` is served here (bxmi.genomic); anything else comes from an installed bx-python (lib/bx/tabular/)."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
