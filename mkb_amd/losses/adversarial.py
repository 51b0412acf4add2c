class Adversarial: pass
