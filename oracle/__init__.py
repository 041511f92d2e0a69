"""CPU oracle for the detect + track hot path (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product (`vehicle-counting_amd/`) never does.  See DESIGN.md section "Oracle".
"""
