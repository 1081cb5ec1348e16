"""Runs the flows of the reference's examples/5_samtron_20D_student-T.py and examples/6_samtron_planar4.py through the
drop-in `gmmvi` alias (same config calls, same GmmviRunner loop), with an iteration cap instead of 1501 iterations /
30 minutes; `stm300` is the shipped 300-dimensional Student-t experiment (configs/experiment_configs/stm300.yml) with the
SAMTRON defaults (sample reuse, adaptive number of components) on the blocked path.
Usage: python tools/run_example_flows.py [stm|planar|stm300] [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmmvi.gmmvi_runner import GmmviRunner
from gmmvi.configs import update_config, get_default_experiment_config, get_default_algorithm_config

which = sys.argv[1] if len(sys.argv) > 1 else "stm"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
algorithm_config = get_default_algorithm_config("SAMTRON")
if which == "stm":
    environment_config = update_config(get_default_experiment_config("stm20"), {"start_seed": 0})
    used = {"num_component_adapter_config": {"del_iters": 100, "add_iters": 60},
            "component_stepsize_adapter_config": {"initial_stepsize": 0.1, "min_stepsize": 0.001, "max_stepsize": 1.},
            "sample_selector_config": {"desired_samples_per_component": 200, "ratio_reused_samples_to_desired": 0.},
            "weight_stepsize_adapter_config": {"initial_stepsize": 1},
            "model_initialization": {"num_initial_components": 45},
            "gmmvi_runner_config": {"log_metrics_interval": 100}}
elif which == "stm300":
    environment_config = update_config(get_default_experiment_config("stm300"), {"start_seed": 0})
    used = {"gmmvi_runner_config": {"log_metrics_interval": 10}}
else:
    environment_config = update_config(get_default_experiment_config("planar_robot_4"), {"start_seed": 0})
    used = {"num_component_adapter_config": {"del_iters": 10, "add_iters": 1},
            "component_stepsize_adapter_config": {"initial_stepsize": 0.1, "min_stepsize": 0.001, "max_stepsize": 1.},
            "sample_selector_config": {"desired_samples_per_component": 100, "ratio_reused_samples_to_desired": 0.},
            "weight_stepsize_adapter_config": {"initial_stepsize": 5},
            "model_initialization": {"num_initial_components": 100},
            "gmmvi_runner_config": {"log_metrics_interval": 50}}
config = update_config(environment_config, update_config(algorithm_config, used))
runner = GmmviRunner.build_from_config(config=config)
t0 = time.time()
elbos = []
for n in range(iters):
    metrics = runner.iterate_and_log(n)
    if "-elbo" in metrics:
        elbos.append(-metrics["-elbo"])
wall = time.time() - t0
print(f"{which}: {iters} iterations in {wall:.1f}s ({iters / wall:.0f} it/s incl. metrics), K = "
      f"{runner.gmmvi.model.num_components}, DB samples = {metrics['num_db_samples']}, ELBO {elbos[0]:.2f} -> {elbos[-1]:.2f}, "
      f"fast path: {runner.gmmvi._fast_path.eligible()}")
assert elbos[-1] > elbos[0]
assert all(e == e for e in elbos)
