"""placeholder (CNN head comes later)"""
MODEL_DOWNSCALE = {"rna004_130bps@v0.2.4.pth": 10}
