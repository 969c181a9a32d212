"""cassierl_amd -- MI355X-native batched Cassie2d environment (hot path of CassieRL/cassierl).

The compute path is the HIP extension lib/libcassie2d.so (csrc/); this package is the
host-side mirror of the reference's Python interface (rllab/envs/cassie2d.py & friends).
There is no CPU fallback: constructing an environment without the extension or without a
HIP device raises.
"""
