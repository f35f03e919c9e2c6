import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import hri_emo_amd
from hri_emo_amd import _lib
v = int(sys.argv[1])
_lib.call("hriemo_rowops_force_variant", v)
import pytest
sys.exit(pytest.main(["tests/test_gpu_parity.py", "-x", "-q", "-k", "test_fusion_train_step_grads", "--tb=line", "-p", "no:cacheprovider"]))
