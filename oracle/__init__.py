"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the RecNet train-step hot path.  Nothing under the product
package may import this; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg do, and only as the checker / the reported CPU baseline.
"""
