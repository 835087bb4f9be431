"""MI355X-backed counterparts of the reference's four hot-path scripts.

Run as ``python -m bxmi.cli.bed_intersect a.bed b.bed`` etc.  Byte-identical stdout;
the work is batched into a handful of kernel launches instead of one call per line."""
