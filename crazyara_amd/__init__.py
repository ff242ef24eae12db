"""crazyara_amd -- MI355X-native batched NN evaluation + MCTS leaf collection behind CrazyAra's NeuralNetAPI surface.

The compute path is the in-tree HIP library (crazyara_amd/lib/libcrazyara_hip.so, built by crazyara_amd.build);
importing the bindings fails loudly when it is missing -- there is no CPU fallback.
"""
__version__ = "0.1.0"
