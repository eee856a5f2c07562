"""CPU oracle for the SEED-X hot path — TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is imported by the product package ``seed-x_amd/``. Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may use it, and only as the checker.

Parity status: the reference ships NO golden vectors / known-answer tests for this path, so
the oracle is pinned against the reference's own modules executed on CPU in the build container
(``oracle/gen_golden.py`` → ``tests/golden/*.npz``, ``tests/test_oracle_vs_reference.py``). The SDXL UNet,
Euler scheduler and t2i CFG loop live in third-party ``diffusers==0.25.0`` (requirements.txt:4), absent from
/root/reference and from this image: their restatement in ``oracle/restated_unet.py`` is **parity unpinned**
(checked only by the exact parameter count 2 567 463 684 and state-dict key layout).
"""
