"""CPU oracle for the dgnn hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a plain PyTorch-CPU / NumPy restatement of the one hot path of
raphaelsulzer/dgnn that `dgnn_amd` accelerates:

* ``learning/surfaceNetStaticEdgeFilters.py``  (SAGEConv :20-109, SurfaceNet :114-355)
* ``learning/surfaceNetUpdatedEdgeFilters.py`` (SAGEConv :23-185, SurfaceNet :188-251)

plus the semantics of the un-vendored third-party pieces those files call
(PyG 2.0.2 ``MessagePassing.propagate``, torch_scatter 2.0.9 ``scatter(reduce='mean')``,
PyG ``BatchNorm``, PyG ``NeighborSampler``), restated in ``pyg_semantics.py``.

Who may import it: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- always as the checker / the timed CPU
baseline, never as the thing shipped.  ``dgnn_amd`` never imports ``oracle`` and
has no CPU fallback: without the HIP extension and a GPU it raises.

Parity pin status
-----------------
The reference ships no tests, golden vectors or fixtures for this path
(SURVEY.md section 4), and the arithmetic lives in un-vendored dependencies.  The
oracle is therefore pinned against *outputs of the reference's own model files
run in the authoring container*: ``tests/golden/make_golden.py`` imports
``/root/reference/learning/surfaceNet{Static,Updated}EdgeFilters.py`` unmodified
(under a small stand-in for the absent ``torch_geometric``/``torch_sparse``
packages), loads the shipped checkpoint ``data/models/kf96/model_best.ptm`` and
writes the ``tests/golden/*.npz`` vectors which ``tests/test_oracle_golden.py``
checks this restatement against (bit-exact in fp32).  The stand-in's semantics
for the PyG/torch_scatter calls are our stated assumption,
cross-checked by an fp64 evaluation stored in the same fixtures.
"""
