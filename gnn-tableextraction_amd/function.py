"""Message / reduce descriptors with the names the reference imports from ``dgl.function``
(src/components/graphs/models.py:3,53-56,148): ``import ... function as fn`` keeps working."""


def u_mul_e(u, e, m):
    return ("u_mul_e", u, e, m)


def copy_u(u, m):
    return ("copy_u", u, m)


def sum(msg, out):  # noqa: A001 - DGL's name
    return ("sum", msg, out)


def mean(msg, out):
    return ("mean", msg, out)
