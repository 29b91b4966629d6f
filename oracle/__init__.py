"""CPU oracle (TEST INFRASTRUCTURE ONLY): importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never from vulkanhybridrenderer_amd/."""
