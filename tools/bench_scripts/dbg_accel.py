import sys; sys.path.insert(0,'.')
import numpy as np
from dither_pie_amd import backend
from dither_pie_amd.dithering_lib import prepare_palette, ColorReducer
for name,pal in [("uniform16",ColorReducer.generate_uniform_palette(16)),("random16",[tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0,256,(16,3))]),("random32",[tuple(int(v) for v in c) for c in np.random.RandomState(7).randint(0,256,(32,3))]),("uniform8",ColorReducer.generate_uniform_palette(8)),("uniform27",ColorReducer.generate_uniform_palette(27))]:
    print(name, file=sys.stderr); P=backend.Palette(*prepare_palette(pal,False), accel=True)
