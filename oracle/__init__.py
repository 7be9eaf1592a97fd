"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the Metadata-Augmented U-Net hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker (never as the thing measured or shipped).

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
(``/root/reference/src/model.py``) in the build container, runs it on seeded
inputs and commits inputs + expected outputs as fixtures under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function of this oracle against
those fixtures (bit-exact where the same torch CPU operators are used).
"""
