"""CPU oracle for the COIN adaptation-training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``coin_amd/`` (the product) may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and there only as the checker / the
reported CPU baseline, never as the thing that is measured or shipped.

Layout
------
``oracle/d2.py``    restatement of the third-party arithmetic the reference
                    calls but does not vendor (detectron2 0.5, torchvision
                    0.10.1 ``roi_align``/``nms``, fvcore ``smooth_l1_loss``).
                    PARITY UNPINNED against upstream (sources absent, no
                    network); pinned by hand-computed known-answer tests in
                    ``tests/test_oracle_d2.py``.
``oracle/coin.py``  restatement of the reference's own modules (file:line cited
                    per function).  Pinned against golden vectors captured by
                    running the reference's modules in the build container
                    (``tests/golden/gen_golden.py``).
"""
