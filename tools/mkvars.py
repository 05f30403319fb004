import sys
from clsurvey_amd import build as b
src = sys.argv[1]
for name in sys.argv[2:]:
    defs = ["CLHIP_ABL_" + d.upper() for d in name.split("+")]
    print(b.build_variant(name.replace("+", "_"), [src], defs, verbose=False))
