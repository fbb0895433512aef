"""graphqembed_amd — MI355X-native (gfx950) conjunctive-query embedding hot path.

Python surface = the reference's (``netquery``): ``graph.Formula/Query/Graph``,
``model.QueryEncoderDecoder``, ``utils.get_*`` / ``eval_*``, ``train_helpers.run_train``.
Arithmetic = hand-written HIP kernels behind the C ABI in ``include/gqe.h``
(``graphqembed_amd/libgqe.so``, built by ``__graft_entry__.build()``).
"""
__version__ = "0.1.0"
