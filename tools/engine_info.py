import importlib, os, sys
os.environ["STCSP_DEBUG"] = "1"
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
for name in sys.argv[1:]:
    print(name, file=sys.stderr)
    e = st.Engine(st.Model.from_name(name)) if not name.startswith("synth") else st.Engine(st.Model(text=st.instances.synthetic(64, 32, 602, 6, 20261003)))
