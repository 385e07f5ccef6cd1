def resnet101(*a, **k):
    raise NotImplementedError("torchvision stub")
