class runtime:
    @staticmethod
    def runtimeGetVersion():
        import torch
        v = getattr(torch.version, "hip", None) or "0.0"
        p = (v.split(".") + ["0", "0"])[:2]
        return int(p[0]) * 1000 + int(p[1]) * 10

    @staticmethod
    def getDeviceCount():
        import torch
        return torch.cuda.device_count()
