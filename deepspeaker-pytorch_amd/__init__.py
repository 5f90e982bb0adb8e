"""deepspeaker-pytorch_amd -- MI355X-native Deep Speaker embedding path.

Drop-in for the model / loss surface of qqueing/DeepSpeaker-pytorch (`model.py`):
`DeepSpeakerModel`, `PairwiseDistance`, `TripletMarginLoss`, plus the triplet filter of
`train_triplet.py:251-262`.  All arithmetic runs in hand-written HIP kernels
(libdeepspeaker_hip.so, C ABI in include/deepspeaker_hip.h); PyTorch only provides device
memory, streams and torch.distributed.
"""
__version__ = "0.1.0"
__all__ = ["__version__"]
