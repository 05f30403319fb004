"""Build experimental variants libclhip_<name>.so (selected at run time with CLHIP_LIB=<path>; tuning only):
    PYTHONPATH=. python tools/mkvars.py <source.hip>[+<source2.hip>] name=DEF1,DEF2=val ..."""
import sys
from clsurvey_amd import build as b
src = sys.argv[1]
for spec in sys.argv[2:]:
    name, defs = spec.split("=", 1) if "=" in spec else (spec, "CLHIP_ABL_" + spec.upper())
    print(b.build_variant(name, src.split("+"), defs.split(","), verbose=False))
