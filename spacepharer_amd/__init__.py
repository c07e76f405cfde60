"""MI355X-native implementation of SpacePHARER's spacer -> phage-ORF search hot path (MMseqs2 prefilter + align +
rescorediagonal).  The product is the C-ABI shared library built from spacepharer_amd/csrc (HIP kernels for gfx950 +
C++ host code); this package is only the thin ctypes binding that bench.py and the tests use.  There is no CPU path:
importing works anywhere, but every compute call needs a HIP device and the in-tree library."""
from ._lib import (SearchContext, SearchParams, Hit, Alignment, Rescore, load_library, library_path, build_library,  # noqa: F401
                   module_prefilter, module_align, module_rescorediagonal, module_search, module_host, HOST_MODULES, SpError, FlatDb)
