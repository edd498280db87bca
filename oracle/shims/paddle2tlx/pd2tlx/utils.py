def restore_model_clas(model, arch, urls):
    raise RuntimeError("pretrained Paddle checkpoints cannot be fetched here (no network)")


restore_model_det = restore_model_seg = restore_model_clas
