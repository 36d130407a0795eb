"""PNG plots with the reference's function names (src/utils/visualization.py:6-82); matplotlib is
imported lazily with the Agg backend.  Not part of the accelerated path."""
from __future__ import annotations


def _plt():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    return plt


def plot_stations_vs_pred(coords, true_vals, pred_vals, save_path="station_comparison.png"):
    plt = _plt()
    fig, ax = plt.subplots(1, 2, figsize=(11, 4.5))
    for a, vals, title in ((ax[0], true_vals, "gauge mean"), (ax[1], pred_vals, "prediction mean")):
        sc = a.scatter(coords[:, 1], coords[:, 0], c=vals, cmap="viridis")
        a.invert_yaxis()
        a.set_title(title)
        fig.colorbar(sc, ax=a)
    fig.savefig(save_path, dpi=300, bbox_inches="tight")
    plt.close(fig)


def plot_training_curves(history, save_path="training_curves.png"):
    plt = _plt()
    fig, ax = plt.subplots(2, 2, figsize=(12, 8))
    e = history["epoch"]
    ax[0, 0].plot(e, history["total_loss"]); ax[0, 0].set_title("total loss")
    for k in ("point_loss", "conserve_loss", "smooth_loss", "temporal_loss"):
        ax[0, 1].plot(e, history[k], label=k)
    ax[0, 1].legend(); ax[0, 1].set_title("loss terms")
    ax[1, 0].plot(e, history["rmse"]); ax[1, 0].set_title("station RMSE")
    ax[1, 1].plot(e, history["learning_rate"]); ax[1, 1].set_title("learning rate")
    fig.savefig(save_path, dpi=300, bbox_inches="tight")
    plt.close(fig)
