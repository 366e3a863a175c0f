"""CPU oracle for the RVIP heatmap-regression U-Net training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product
path: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` may import it, and there only as the checker.

PARITY UNPINNED: the reference (Cardio-AI/cmr-landmark-detection) executes this
path entirely inside TensorFlow 2.3 / Keras, which is not importable here, and
it ships no numerical golden vectors for it.  The only reference-owned pin is
the stored ``model.summary()`` printout (tests/golden/model_summary.json), which
``rvip_oracle.build_graph`` reproduces exactly.  Numerics are cross-checked
against an independent implementation (PyTorch-CPU, ``oracle/torch_ref.py``).
"""
