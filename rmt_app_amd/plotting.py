"""`display-result == "True"` hook (reference: plotResultsDynamic,
PyREMOT/solvers/solResultAnalysis.py:373-459).  Plotting is outside the hot path (SURVEY.md
section 8(f) rank 4); this draws the outlet histories with matplotlib if it is importable."""


def plot_results_dynamic(resPack, tNo):
    try:
        import matplotlib.pyplot as plt
    except Exception:  # pragma: no cover
        print("display-result requested but matplotlib is unavailable")
        return
    packs = resPack["dataPack"]
    labels = packs[0]["labelList"]
    fig, axes = plt.subplots(1, 2, figsize=(10, 4))
    for d in packs:
        for i, lab in enumerate(labels[:-1]):
            axes[0].plot(d["dataXs"], d["dataYs"][i], label="%s t=%.3g" % (lab, d["dataTime"]))
        axes[1].plot(d["dataXs"], d["dataYs"][-1], label="t=%.3g s" % d["dataTime"])
    axes[0].set_xlabel("z*"), axes[0].set_ylabel("mole fraction")
    axes[1].set_xlabel("z*"), axes[1].set_ylabel("T [K]"), axes[1].legend()
    plt.show()
