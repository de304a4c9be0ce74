"""face_recognition_amd -- MI355X-native embed-and-match hot path of CEIA-NoLeakIV/face-recognition.

Host-side mirror of the reference's Python entry points (inference.py, evaluate.py) over the C ABI
of libfr_hip.so (include/fr_hip.h).  Import as ``import face_recognition_amd`` (shim at repo root).
"""
from . import _lib
from . import ops
from .models import (HipBackbone, MobileNetV1, MobileNetV2, mobilenet_v3_large, mobilenet_v3_small,
                     sphere20, sphere36, sphere64, default_precision)
from .inference import (get_network, load_model, get_transform, extract_features, extract_batch_embeddings,
                        compare_faces, compute_similarity, embed_images, cosine_pairs)
from . import evaluate
from . import search
from . import gallery
from . import align

__all__ = ["get_network", "load_model", "get_transform", "extract_features", "extract_batch_embeddings",
           "compare_faces", "compute_similarity", "embed_images", "cosine_pairs", "evaluate", "search", "gallery", "align",
           "HipBackbone", "MobileNetV1", "MobileNetV2", "mobilenet_v3_large", "mobilenet_v3_small",
           "sphere20", "sphere36", "sphere64", "default_precision"]
