"""Host-side orchestration with the reference's interfaces (config, dataset, early stopping,
trainer).  Pure PyTorch-ROCm / numpy plumbing around the HIP engine; nothing here is a kernel."""
