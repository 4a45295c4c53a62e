"""aindex_amd — MI355X-native k-mer counting and perfect-hash lookup engine.

Drop-in for the count_kmers / compute_index / compute_aindex / batch get_tf_values path of
ad3002/aindex: same `AindexWrapper` / `AIndex` Python API and the same on-disk `.pf`, `.tf.bin`,
`.kmers.bin`, `.index.bin`, `.indices.bin` layouts; the work runs in hand-written HIP kernels for
gfx950 behind the C ABI declared in include/aindex_hip.h.
"""
__version__ = "0.1.0"
