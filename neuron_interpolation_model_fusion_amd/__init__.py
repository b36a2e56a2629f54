"""MI355X-native neuron matching-and-refit path for model fusion.

Host classes (same names and signatures as the reference's fusion_algorithms) sit on
``ops`` -- the torch-tensor front end of libnif.so (hand-written gfx950 HIP kernels).
"""
__version__ = '0.1.0'
