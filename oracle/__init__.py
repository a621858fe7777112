"""CPU oracle for the MPC controller path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  PARITY UNPINNED for the MPC arithmetic (see mpc_oracle.h header).
"""
