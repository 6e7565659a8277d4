from .resnet_inst_matt_spconv import res_shortcut_inst_matt_spconv_22          # MaGGIe: IMD + sparse refinement
from .resnet_inst_matt_spconv_temp import res_shortcut_inst_matt_spconv_temp_22  # MaGGIe_Temp: + temporal modules
