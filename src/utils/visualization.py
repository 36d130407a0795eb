"""Drop-in import path of the reference (`from src.utils.visualization import ...`)."""
from pl_convlstm_gan_amd.host.visualization import plot_stations_vs_pred, plot_training_curves  # noqa: F401
