"""ptgnn_amd -- MI355X-native message-passing core behind microsoft/ptgnn's layer API.

Only the hot path lives here: graph plan (CSR) construction, the fused
gather/segment-reduce kernel, fp32-MFMA dense blocks, and `nn.Module`s that mirror ptgnn's
`AbstractMessagePassingLayer` / `GraphNeuralNetwork` surface so `GnnOutput` consumers are drop-in.
"""
from ptgnn_amd._lib import PtgnnAmdError  # noqa: F401

__version__ = "0.1.0"
