import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tools"))
from benchlib.legs import e2e_from_files
print(e2e_from_files(32, 3, 0, big_pairs=int(sys.argv[1]) if len(sys.argv) > 1 else 0), flush=True)
