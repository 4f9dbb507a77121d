"""coverm_amd — MI355X-native engine for CoverM's BAM -> pileup -> per-contig / per-genome path.

The product is libcovermhip.so (hand-written gfx950 HIP kernels behind the C ABI of
include/covermhip.h) plus the host side above it.  Importing this package never touches oracle/.
"""
__version__ = "0.1.0"
