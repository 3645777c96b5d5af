import os, sys, subprocess
shapes = "64,64,96,128,3 64,64,56,56,3 256,64,56,56,1 1024,256,14,14,1 256,256,96,128,1"
for abl in (0, 1, 2, 3, 4, 8, 7, 15):
    env = dict(os.environ, VITTA_SK_ABL=str(abl))
    out = subprocess.run([sys.executable, "tools/debug/sk_abl_child.py"] + shapes.split(), env=env, capture_output=True, text=True).stdout
    print("abl", abl, "|", out.strip().replace("\n", " | "), flush=True)
