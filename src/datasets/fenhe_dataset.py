"""Drop-in import path of the reference (`from src.datasets.fenhe_dataset import FenheDataset`)."""
from pl_convlstm_gan_amd.host.datasets import FenheDataset, get_shapefile_extent  # noqa: F401
