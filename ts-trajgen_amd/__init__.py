"""ts-trajgen_amd: MI355X-native A*-guided trajectory generation (the TS-TrajGen hot path).

Host side: Python mirrors of the reference's `GeneratorV4` / `Searcher` call signatures.
Device side: hand-written gfx950 HIP kernels behind the C-ABI of `include/tsg.h`
(`csrc/libtsg.so`).  There is no CPU fallback: without the built library every compute
entry point raises.
"""
__version__ = '0.1.0'
