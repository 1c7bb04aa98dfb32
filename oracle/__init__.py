"""CPU oracle for the render-and-train hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``nerf_amd/`` imports this package; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do,
and there only as the checker / the timed CPU baseline, never as the product path.
"""
