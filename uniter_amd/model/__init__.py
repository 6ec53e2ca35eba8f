"""Drop-in counterparts of the reference's ``model`` package (model/model.py, layer.py, attention.py,
pretrain.py, vqa.py, nlvr2.py): same class names, constructor signatures, attribute paths and
state_dict keys; the encoder runs on the HIP kernels of libuniter_hip.so."""
from .model import (UniterConfig, UniterEncoder, UniterImageEmbeddings, UniterModel,  # noqa: F401
                    UniterPreTrainedModel, UniterTextEmbeddings)
