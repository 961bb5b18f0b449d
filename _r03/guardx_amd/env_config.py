"""Task name -> Engine config dict, and `create_env(args)`.

Counterpart of safe_rl_libX/guard_utils/safe_rl_env_config.py for the task
family the batched Engine can run at the reference commit (SURVEY.md fact 7:
only `Goal_<Robot>_8Hazards`; the other ~85 branches carry keys the reference
Engine itself rejects with "Bad key").
"""


def _goal_task(robot_xml, extra=None):
    cfg = {
        'robot_base': robot_xml,
        'task': 'goal',
        'goal_size': 0.5,
        'observe_goal_comp': True,
        'observe_hazards': True,
        'constrain_hazards': True,
        'constrain_indicator': False,
        'lidar_num_bins': 16,
        'hazards_num': 8,
        'hazards_size': 0.3,
    }
    if extra:
        cfg.update(extra)
    return cfg


def configuration_list(task):
    """safe_rl_env_config.py:6-2568 restricted to the runnable Goal family."""
    if task == "Goal_Point_8Hazards":        # safe_rl_env_config.py:59-81
        return _goal_task('xmls/point.xml')
    if task == "Goal_Swimmer_8Hazards":      # :111-134
        return _goal_task('xmls/swimmer.xml', {
            'sensors_obs': ['accelerometer', 'velocimeter', 'gyro', 'magnetometer',
                            'touch_point1', 'touch_point2', 'touch_point3', 'touch_point4']})
    if task == "Goal_Ant_8Hazards":          # :195-220
        return _goal_task('xmls/ant.xml', {
            'sensors_obs': ['accelerometer', 'velocimeter', 'gyro', 'magnetometer',
                            'touch_ankle_1a', 'touch_ankle_2a', 'touch_ankle_3a', 'touch_ankle_4a',
                            'touch_ankle_1b', 'touch_ankle_2b', 'touch_ankle_3b', 'touch_ankle_4b']})
    if task == "Goal_Walker_8Hazards":       # :254-279
        return _goal_task('xmls/walker.xml', {
            'sensors_obs': ['accelerometer', 'velocimeter', 'gyro', 'magnetometer',
                            'touch_right_foot', 'touch_left_foot']})
    if task == "Goal_Doggo_8Hazards":        # :167-193 -- the Engine refuses it: no HIP dynamics for xmls/doggo.xml
        return _goal_task('xmls/doggo.xml', {
            'sensors_obs': ['accelerometer', 'velocimeter', 'gyro', 'magnetometer',
                            'touch_ankle_1a', 'touch_ankle_2a', 'touch_ankle_3a', 'touch_ankle_4a',
                            'touch_ankle_1b', 'touch_ankle_2b', 'touch_ankle_3b', 'touch_ankle_4b']})
    if task == "Ant_8Hazards_8Pillars_synthetic":
        # BASELINE.json config 5 ("Push_Ant_8Hazards+8Pillars").  NO REFERENCE COUNTERPART: the reference's
        # Push_Ant_8Hazards (:768-796) carries 'observe_box_comp', which Engine.parse rejects (engine.py:326-328),
        # the push task places no goal (engine.py:538) and pillars exist only as two constants (engine.py:38,56).
        # Synthetic stand-in per SURVEY.md section 8d: ant.xml (foot-floor contacts), the goal task, 8 hazards and 8
        # static pillar circles in the hazard style (own lidar, keepout .3, size .2; guardx_amd.Engine.EXTENSIONS).
        # The arena is 6 m x 6 m: 18 objects with their keepouts do not fit the reference's 4 m x 4 m in 10 tries.
        return _goal_task('xmls/ant.xml', {
            'pillars_num': 8, 'observe_pillars': True, 'pillars_keepout': 0.3, 'pillars_size': 0.2,
            'placements_extents': [-3, -3, 3, 3]})
    return {}  # unknown names fall through to Engine defaults, as in the reference


def configuration(task):
    """safe_rl_env_config.py:2570-2595 (configuration_list never raises, so the
    name-parsing fallback there is unreachable)."""
    return configuration_list(task)


def create_env(args, **engine_kwargs):
    """safe_rl_env_config.py:2597-2614.  `args` needs .task .env_num .seed .max_ep_len"""
    from .engine import Engine
    config = configuration(args.task)
    config['env_num'] = args.env_num
    config['_seed'] = args.seed
    config['num_steps'] = args.max_ep_len
    config['device_id'] = getattr(args, 'device_id', 0)
    return Engine(config, **engine_kwargs)
