"""MI355X-native denoising hot path of keypoint-diffusion.

Public surface mirrors the reference modules:
    KeypointDiffusion (alias LigandDiffuser), LigRecDynamics, LigRecDynamicsGVP,
    ReceptorEncoderGVP, FixedReceptorEncoder, model_from_config, and the graph container
    that stands in for the DGL heterograph.
"""
__version__ = '0.1.0'
