"""Drop-in import path of the reference (`from src.datasets.fenhe_dataset_split import ...`)."""
from pl_convlstm_gan_amd.host.datasets import (FenheDataset, get_shapefile_extent,  # noqa: F401
                                               split_dataset_by_year, split_dataset_random)
