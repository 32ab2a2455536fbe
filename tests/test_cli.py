"""Flag surface of the command line (reference lungmask/__main__.py:26-76) -- no GPU needed for parsing."""
import numpy as np
import pytest

from lungmask_amd.__main__ import build_parser


def test_cli_flags_match_reference(tmp_path):
    f = tmp_path / "v.npy"
    np.save(f, np.zeros((1, 16, 16), np.int16))
    a = build_parser().parse_args([str(f), "out.npy"])
    assert (a.modelname, a.modelpath, a.cpu, a.nopostprocess, a.batchsize, a.noprogress, a.removemetadata) == ("R231", None, False, False, 20, False, False)
    a = build_parser().parse_args([str(f), "o.npy", "--modelname", "LTRCLobes_R231", "--batchsize", "5", "--nopostprocess", "--noprogress", "--removemetadata"])
    assert a.modelname == "LTRCLobes_R231" and a.batchsize == 5 and a.nopostprocess and a.noprogress and a.removemetadata
    with pytest.raises(SystemExit):
        build_parser().parse_args([str(f), "o.npy", "--modelname", "nope"])
    with pytest.raises(SystemExit):
        build_parser().parse_args([str(tmp_path / "missing.npy"), "o.npy"])  # "File not found"
