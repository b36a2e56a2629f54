"""CPU oracle: a restatement of the reference's per-level matching-and-refit arithmetic.

TEST INFRASTRUCTURE ONLY.  Nothing under ``neuron_interpolation_model_fusion_amd`` may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and only as the checker / the timed CPU baseline.

Every function cites the reference file:line it follows and replays the reference's own op
sequence on the CPU (torch.linalg.pinv, the chunk-8 cost loop, full-recompute farthest-point
init, Python centroid loops, scipy.optimize.linear_sum_assignment), so it doubles as the CPU
baseline.  Pinned by tests/golden/*.npz, minted from the imported reference by
tests/golden/make_golden.py (GR, HF, KF, ALF k-means).  The OTFusion restatement is
PARITY-UNPINNED at ``ot.emd``: POT is not installed, fusion_algorithms/otfusion.py cannot be
imported, and the reference holds no fixture for it; ``emd`` below is an exact transport
solve through scipy instead (linear_sum_assignment for uniform square problems, HiGHS LP
otherwise).
"""
