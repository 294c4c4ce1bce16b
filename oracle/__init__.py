"""TEST INFRASTRUCTURE ONLY — CPU oracle for the IC-GAN G+D hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the *checker*.  The product path (``ic_gan_amd``) never
imports this package and fails loudly when its HIP library is missing.

Parity status: PINNED.  ``oracle.biggan_oracle`` is checked against outputs of
the unmodified reference modules run on CPU in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; SURVEY.md §8c: the
reference ships no tests or golden vectors of its own, so reference outputs
generated here are the contract).
"""
