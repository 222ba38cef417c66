"""pushworld_amd -- MI355X-native batched PushWorld step engine.

Drop-in for the hot path of google-deepmind/pushworld (``PushWorldPuzzle.get_next_state`` /
``render`` behind ``PushWorldEnv.step``): hand-written gfx950 HIP kernels behind a C ABI
(``include/pushworld_amd.h``), exposed through the reference's own Python surface.

    from pushworld_amd.puzzle import PushWorldPuzzle, Actions      # pushworld.puzzle
    from pushworld_amd.gym_env import PushWorldEnv                 # pushworld.gym_env
    from pushworld_amd.dm_env import PushWorldEnv as DmEnv         # pushworld.dm_env
    from pushworld_amd.vec_env import VecPushWorld                 # batched, new
    from pushworld_amd.vector_env import PushWorldVectorEnv        # gymnasium.vector surface, new
"""
__version__ = "0.1.0"
