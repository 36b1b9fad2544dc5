"""Result display / export layer against golden G10, which
tools/make_golden.py `plot` records from the reference's own functions (plotResultsDynamic,
plotResultsSteadyState with plots2D intercepted; selectRandomForList under fixed numpy seeds;
saveResultClass file contents)."""
import json
import os

import numpy as np

from rmt_app_amd import plotting as PL
from rmt_app_amd.save_result import saveResultClass as sRes

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def respack():
    S, N, tNo = 3, 6, 5
    xs = np.linspace(0, 1, N)
    packs = []
    for k in range(tNo):
        ys = np.array([[0.1*(i + 1) + 0.01*k + 0.001*j for j in range(N)] for i in range(S)] +
                      [[500.0 + k + 0.5*j for j in range(N)]])
        packs.append({"modelId": "N2", "processType": "non-iso-thermal", "successStatus": True,
                      "dataShape": (S + 1, N), "labelList": ["A", "B", "C", "Temperature"],
                      "indexList": [S, S + 1, S], "dataTime": 0.1*(k + 1), "dataXs": xs, "dataYs": ys})
    return {"computation-time": 1.234, "dataPack": packs}, tNo


def same_figures(got, want):
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert (a["title"], a["xlabel"], a["ylabel"]) == (b["title"], b["xlabel"], b["ylabel"])
        assert [l["leg"] for l in a["lines"]] == [l["leg"] for l in b["lines"]]
        for la, lb in zip(a["lines"], b["lines"]):
            assert np.array_equal(np.asarray(la["x"]), np.asarray(lb["x"]))
            assert np.array_equal(np.asarray(la["y"]), np.asarray(lb["y"]))


def test_random_time_slices_match_reference():
    g = json.load(open(os.path.join(G, "g10_plot_export.json")))
    for seed, want in g["picks"].items():
        np.random.seed(int(seed))
        assert [int(v) for v in PL.selectRandomForList(list(range(5)), 2)] == want
    np.random.seed(3)
    assert [int(v) for v in PL.selectRandomForList(list(range(10)), 2)] == g["picks10"]
    assert PL.selectFromListByIndex([], [1, 2, 3]) == g["select"][0]
    assert PL.selectFromListByIndex([2, 0], [1, 2, 3]) == g["select"][1]


def test_plot_results_dynamic_and_steady_state_match_reference():
    g = json.load(open(os.path.join(G, "g10_plot_export.json")))
    rp, tNo = respack()
    np.random.seed(11)
    same_figures(PL.plotResultsDynamic(rp, tNo, show=False), g["dynamic"])
    d = dict(rp["dataPack"][0])
    d.update({"modelId": "N1", "computation-time": 0.5, "labelList": ["A", "B", "C", "Pressure", "Temperature"],
              "indexList": [3, 3, 4], "dataYs": np.vstack([d["dataYs"][:3], np.linspace(50, 49, 6), d["dataYs"][3:]])})
    same_figures(PL.plotResultsSteadyState([d], show=False), g["steady"])
    # iso-thermal packs have no temperature figure (solResultAnalysis.py:443-444)
    for p in rp["dataPack"]:
        p["processType"] = "iso-thermal"
    np.random.seed(11)
    assert len(PL.plotResultsDynamic(rp, tNo, show=False)) == 4


def test_save_result_files_match_reference(tmp_path, capsys):
    g = json.load(open(os.path.join(G, "g10_plot_export.json")))
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        sRes.saveListToText([1.5, "abc", [1, 2], np.float64(2.25)])
        sRes.saveListToCSV([[1, 2.5, "x"], [3, 4.0, "y,z"]], ["a", "b", "c"])
        assert open("saveFile.txt", newline="").read() == g["txt"]
        assert open("saveFile.csv", newline="").read() == g["csv"]
        sRes.saveListToText("not a list")
        assert "data is not a list" in capsys.readouterr().out
    finally:
        os.chdir(cwd)
