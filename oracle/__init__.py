"""CPU oracle for the TexIR hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Nothing under texir_code_amd/ does.
"""
