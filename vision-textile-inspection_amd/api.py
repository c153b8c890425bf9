"""Public names of the vti_amd package."""
from ._lib import LIB_PATH, SIGNATURES, VtiError, lib
from .engine import Engine, debug_conv2d, h2_decode, h2_encode, kmeans1d2, pixels_to_world, unpack_bits
from .feeder import FrameFeeder
from .model import YOLO, Boxes, Masks, Results, letterbox_shape
from .weights import pack_container, random_weights, unpack_container
from .convert import convert_checkpoint, convert_state_dict
from . import consumer, dataparallel

__all__ = ["LIB_PATH", "SIGNATURES", "VtiError", "lib", "Engine", "debug_conv2d", "h2_decode", "h2_encode", "kmeans1d2", "pixels_to_world", "unpack_bits", "FrameFeeder", "YOLO", "Boxes", "Masks",
           "Results", "letterbox_shape", "pack_container", "random_weights", "unpack_container", "convert_checkpoint", "convert_state_dict",
           "consumer", "dataparallel"]
