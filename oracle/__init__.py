"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the SeedVR2 hot path (NaDiT forward + causal-Conv3d video
VAE encode/decode) used as the parity checker for the HIP product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import anything from here, and only as the checker / the reported CPU
baseline -- never as the thing that is measured or shipped.  The product
package (``comfyui-seedvr2_videoupscaler_amd``) never imports ``oracle``.

Parity pinning: the reference repo ships NO tests, golden vectors or KATs for
this path, so the restatement is pinned against outputs
of the reference implementation itself, imported unmodified from
``/root/reference`` through ``oracle/third_party_shims.py`` by
``oracle/make_golden.py`` (committed fixtures under ``tests/golden/``), and,
whenever ``/root/reference`` is mounted, live in ``tests/test_oracle_vs_reference.py``.
"""
