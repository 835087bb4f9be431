"""bx.bitset_builders -- lib/bx/bitset_builders.py:17-157's builders on device bitsets (bxmi.builders)."""
from bx.bitset import MAX, BinnedBitSet  # noqa: F401  (names the reference module has too)
from bxmi.builders import (  # noqa: F401
    binned_bitsets_by_chrom,
    binned_bitsets_from_bed_file,
    binned_bitsets_from_file,
    binned_bitsets_from_list,
    binned_bitsets_proximity,
)
