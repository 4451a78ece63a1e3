"""CPU oracle of the PicoPose hot path — TEST INFRASTRUCTURE ONLY.

A restatement of the reference's algorithm used to check the HIP path.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; nothing under picopose_amd/ does.
"""
