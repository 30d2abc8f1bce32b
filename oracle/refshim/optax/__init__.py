"""optax stand-in: imported at module level by the reference's trainer module; the LOSS functions executed here
(trainers/proj/image_text/_deprecated_contrastive.py:80-200) do not touch it.  The optimizer chain is pinned elsewhere
(tests/test_oracle.py: the reference's own known answers, optax_test.py:103-318)."""
