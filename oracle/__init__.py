"""CPU oracle for the ALS hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (implicit_amd) never does: it fails loudly without its HIP library.
"""
