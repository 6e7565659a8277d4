from .maggie import MaGGIe
from .maggie_temp import MaGGIe_Temp
