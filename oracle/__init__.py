"""CPU oracle for the cWGAN-GP hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / reported baseline.  The product path
(``pr_disagg_radar_gan_amd``) never imports it and fails loudly without the HIP library.

PARITY UNPINNED: the reference (sipposip/pr-disagg-radar-gan) is Python on TensorFlow 2.1
(gan_train_cwgangp_pixelnorm.py:37), TensorFlow is not installed here, its pretrained
``trained_models/*.h5`` blobs are absent (.MISSING_LARGE_BLOBS:3-4) and the reference has no
tests, golden vectors or fixtures for this path.  What pins the oracle instead: two
independent restatements written from the op definitions (numpy fp64 loops over taps in
``rdgan_np``; torch-CPU with autograd in ``rdgan_torch``) that must agree, fp64 finite
differences for every gradient incl. the gradient-penalty double backward, and the
hand-derivable known-answer tests of SURVEY.md section 8c (tests/test_oracle_kat.py).
"""
