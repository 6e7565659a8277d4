from .spectral_norm import SpectralNorm
from .base import conv1x1, conv3x3, ConvWeight, Marker
from .aspp import ASPP
from .conv_gru import ConvGRU
from .instance_matte_decoder import InstanceMatteDecoder
