"""Building blocks of the MI355X MaGGIe build. Exported under the names the reference's `maggie.network.module` uses, so that
code written against it (`from maggie.network.module import SpectralNorm, ASPP, ...`) resolves here too."""
from . import aspp, base, conv_gru, instance_matte_decoder, spectral_norm

ASPP = aspp.ASPP
ConvGRU = conv_gru.ConvGRU
InstanceMatteDecoder = instance_matte_decoder.InstanceMatteDecoder
SpectralNorm = spectral_norm.SpectralNorm
ConvWeight, Marker, conv1x1, conv3x3 = base.ConvWeight, base.Marker, base.conv1x1, base.conv3x3

__all__ = ['ASPP', 'ConvGRU', 'ConvWeight', 'InstanceMatteDecoder', 'Marker', 'SpectralNorm', 'conv1x1', 'conv3x3']
