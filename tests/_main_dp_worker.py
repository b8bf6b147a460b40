"""Worker of tests/test_gpu_main.py::test_cli_two_ranks_reproduce_the_reference_run: one of two data-parallel ranks of the
training CLI (both on cuda:0, gloo rendezvous).  Writes what it observed to <odir>/rank<r>.json."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch
    import main_fixture
    from superpoint_graph_amd.learning import datasets, main as cli
    odir, extra = sys.argv[1], sys.argv[2:]
    train, test = main_fixture.make_dataset(0)
    points = {name: pts for name, _, pts in train + test}
    datasets.register_memory_dataset('memtest', [(n, g) for n, g, _ in train], [(n, g) for n, g, _ in test], [], points,
                                     main_fixture.N_CLASSES, 14)
    session = cli.main(['--dataset', 'memtest', '--odir', odir] + main_fixture.CLI +
                       ['--dist_backend', 'gloo', '--dist_device', '0'] + extra)
    torch.cuda.synchronize()
    out = {'rank': session.rank, 'world': session.world,
           'train_losses': [float(x[0]) for x in session.iter_log], 'eval_losses': [float(x) for x in session.eval_log],
           'stats': session.stats,
           'param_l2': {k: float(v.double().norm()) for k, v in session.model.state_dict().items() if v.is_floating_point()}}
    with open(os.path.join(odir, 'rank%d.json' % session.rank), 'w') as f:
        json.dump(out, f)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
