"""Result display layer: what happens after the integration when
``solver-config.display-result == "True"``.  Host-side only; nothing here touches the device.

  * ``plots2DSetXYList`` / ``plots2DSetDataList`` / ``plots2D``  <- PyREMOT/library/plot.py:30-115
  * ``selectFromListByIndex`` / ``selectRandomForList``          <- PyREMOT/core/utilities.py:25-39, 88-109
  * ``plotResultsDynamic(resPack, tNo)``                         <- PyREMOT/solvers/solResultAnalysis.py:373-459
  * ``plotResultsSteadyState(dataPack)``                         <- PyREMOT/solvers/solResultAnalysis.py:307-368

Like the reference, the dynamic plot shows the first and the last output time plus TWO interior
ones drawn with ``numpy.random.choice`` (so a seeded ``numpy.random`` gives the reference's picks).
Each function returns the figures it drew as a list of ``{"title", "xlabel", "ylabel", "lines"}``
dicts, which is what the tests compare; drawing needs matplotlib and is skipped when it is missing.
"""
import numpy as np

from .settings import PROCESS_SETTING


def plots2DSetXYList(X, Ys):
    """[[X, y] for every row y]  (plot.py:85-90)"""
    return [[X, item] for item in Ys]


def plots2DSetDataList(XYList, labelList):
    """[{x, y, leg}]  (plot.py:93-115)"""
    return [{"x": XYList[i][0], "y": XYList[i][1], "leg": labelList[i]} for i in range(len(XYList))]


def selectFromListByIndex(indices, refList):
    """utilities.py:25-39: an empty index list selects everything"""
    return [refList[index] for index in indices] if len(indices) != 0 else refList


def selectRandomForList(myList, no):
    """first element, ``no`` sorted random interior INDICES, last element (utilities.py:88-109)"""
    idx = [*range(np.shape(myList)[0])][1:-1]
    picked = np.sort(np.random.choice(idx, no, replace=False))
    return [myList[0], *picked, myList[-1]]


def plots2D(data, xLabel, yLabel, title="", show=True):
    """One figure with a line per entry of ``data`` (plot.py:30-82)."""
    lines = data if isinstance(data, list) else [data]
    fig = {"title": title, "xlabel": xLabel, "ylabel": yLabel,
           "lines": [{"x": d["x"], "y": d["y"], "leg": d.get("leg", "line")} for d in lines]}
    if show:
        try:
            import matplotlib.pyplot as plt
        except Exception:  # pragma: no cover
            print("display-result requested but matplotlib is unavailable")
            return fig
        for d in fig["lines"]:
            plt.plot(d["x"], d["y"], label=d["leg"])
        if len(title) > 0:
            plt.title(title)
        plt.xlabel(xLabel)
        plt.ylabel(yLabel)
        plt.legend()
        plt.show()
    return fig


def plotResultsDynamic(resPack, tNo, show=True):
    elapsed = resPack['computation-time']
    dataPack = resPack['dataPack']
    modelId = dataPack[0]['modelId']
    processType = dataPack[0]['processType']
    labelList = dataPack[0]['labelList']
    indexList = dataPack[0]['indexList']
    plotTitle = f"Steady-State Modeling {modelId}, computation-time {elapsed}"     # sic (:413)
    xLabelSet = "Reactor Length (m)"
    yLabelSet = ("Concentration (mol/$m^3$)", "Temperature (K)")
    compNo, indexTemp = indexList[0], indexList[2]
    figures = []
    for i in selectRandomForList(list(range(tNo)), 2):                             # :421-422
        d = dataPack[i]
        if d['successStatus'] is not True:
            break                                                                   # :456-457
        title = plotTitle + f" at t={d['dataTime']}"
        dataList = plots2DSetDataList(plots2DSetXYList(d['dataXs'], d['dataYs']), labelList)
        dataLists = ([dataList[0:compNo], dataList[indexTemp]]
                     if processType != PROCESS_SETTING['ISO-THER'] else [dataList[0:compNo]])
        for f, sel in enumerate(selectFromListByIndex([], dataLists)):
            figures.append(plots2D(sel, xLabelSet, yLabelSet[f], title, show))
    return figures


def plotResultsSteadyState(dataPack, show=True):
    d = dataPack[0]
    plotTitle = f"Steady-State Modeling {d['modelId']}, computation-time {d['computation-time']}"
    xLabelSet = "Reactor Length (m)"
    yLabelSet = ("Concentration (mol/$m^3$)", "Pressure (bar)", "Temperature (K)")
    compNo, indexPressure, indexTemp = d['indexList'][0], d['indexList'][1], d['indexList'][2]
    figures = []
    if d['successStatus'] is True:
        dataList = plots2DSetDataList(plots2DSetXYList(d['dataXs'], d['dataYs']), d['labelList'])
        dataLists = ([dataList[0:compNo], dataList[indexPressure], dataList[indexTemp]]
                     if d['processType'] != PROCESS_SETTING['ISO-THER']
                     else [dataList[0:compNo], dataList[indexPressure]])
        for f, sel in enumerate(selectFromListByIndex([], dataLists)):
            figures.append(plots2D(sel, xLabelSet, yLabelSet[f], plotTitle, show))
    return figures


# name used by run_n2 since the first round
plot_results_dynamic = plotResultsDynamic
