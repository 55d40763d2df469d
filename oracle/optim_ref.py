"""CPU restatement of the optimiser step on the training path -- TEST INFRASTRUCTURE ONLY (only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import oracle/).

Adam as shipped in the reference's torch 0.4.1 (/root/reference/venv_vid2vid/lib/python3.7/site-packages/torch/optim/
adam.py:48-98), which vid2vid runs with lr 2e-4, betas (0.5, 0.999).  Pinned: tests/golden/
adam041.npz holds parameters produced by that very file (tests/golden/make_adam_golden.py imports it from where it
lies); tests/test_cpu_oracle_and_host.py checks this restatement against them.
"""


def adam_041_step(p, g, m, v, lr, b1, b2, eps, step):
    """One in-place Adam step on tensors of any float dtype (adam.py:90-98).  `eps` is added to sqrt(v) BEFORE the
    bias corrections are folded into the step size -- modern torch divides sqrt(v) by sqrt(bias_correction2) first."""
    m.mul_(b1).add_(g, alpha=1 - b1)                         # exp_avg.mul_(beta1).add_(1 - beta1, grad)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)                  # exp_avg_sq.mul_(beta2).addcmul_(1 - beta2, grad, grad)
    denom = v.sqrt().add_(eps)                               # denom = exp_avg_sq.sqrt().add_(group['eps'])
    step_size = lr * (1 - b2 ** step) ** 0.5 / (1 - b1 ** step)
    p.addcdiv_(m, denom, value=-step_size)                   # p.data.addcdiv_(-step_size, exp_avg, denom)
