//! Safe Rust wrapper over the MI355X-native GSWT hot path (`libgswt_hip.so`, `libgswt_host.so`).
//!
//! SOURCE ONLY — the build image of this repository has no Rust toolchain; this file has never been compiled there.
//! The `*_sys` modules are generated from `include/*.h`; their layouts are verified against gcc by
//! `tests/test_abi_symbols.py`.
//!
//! The types keep the reference's names so that `state.rs` changes by an import line:
//!
//! | reference (`zengyf131/gswt_renderer`)                               | here                                   |
//! |---------------------------------------------------------------------|----------------------------------------|
//! | `renderer::GSWTRenderer::new(.., PreloadData)` (`renderer.rs:31`)    | [`GSWTRenderer::new`]                  |
//! | `GSWTRenderer::configure(.., &UserData, ..)` (`renderer.rs:351`)     | [`GSWTRenderer::configure`]            |
//! | swap-in of a `SortData` (`state.rs:361-376`)                         | [`GSWTRenderer::set_sort_data`]        |
//! | `GSWTRenderer::render(.., &Camera, &RenderData)` (`renderer.rs:407`) | [`GSWTRenderer::render`]               |
//! | `scene::load_scene_zip` (`scene.rs:1030`)                            | [`WangTile::from_zip`]                 |
//! | `WangTile::{configure, check_update, build_tiles, sort_tiles}`       | the methods of [`WangTile`]            |
pub mod gswt_hip_sys;
pub mod gswt_host_sys;

use gswt_hip_sys as hip;
use gswt_host_sys as host;
use std::ffi::{CStr, CString};
use std::os::raw::c_int;
use std::ptr;

/// Status of a failed call + the library's message (`gswt_last_error` / `gswt_host_last_error`).
#[derive(Debug, Clone)]
pub struct GswtError {
    pub code: c_int,
    pub message: String,
}

impl std::fmt::Display for GswtError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "gswt error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for GswtError {}

pub type Result<T> = std::result::Result<T, GswtError>;

fn host_check(rc: c_int) -> Result<()> {
    if rc == hip::GSWT_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(host::gswt_host_last_error()) };
    Err(GswtError { code: rc, message: msg.to_string_lossy().into_owned() })
}

/// `wangtile::WangTile` on `libgswt_host.so` (C++ mirror of the reference's worker; one owner thread).
pub struct WangTile {
    raw: *mut host::GswtWang,
    user: Option<host::GswtUserData>,
    conf: Option<host::GswtConfigured>,
    scene: Option<host::GswtSceneData>,
}

unsafe impl Send for WangTile {} // moved to the worker thread exactly like the reference's WangTile (state.rs:440,478)

impl WangTile {
    /// `load_scene_zip(path).await` + `WangTile::new(scene_vec)` (`state.rs:~125`, `wangtile.rs:41`).
    pub fn from_zip(path: &str) -> Result<Self> {
        let cpath = CString::new(path).map_err(|_| GswtError { code: hip::GSWT_ERR_BAD_ARG, message: "path holds a NUL".into() })?;
        let mut ts: *mut host::GswtTileset = ptr::null_mut();
        host_check(unsafe { host::gswt_load_scene_zip(cpath.as_ptr(), &mut ts) })?;
        let mut raw: *mut host::GswtWang = ptr::null_mut();
        host_check(unsafe { host::gswt_wang_new(ts, &mut raw) })?; // takes the tile set, also on failure
        Ok(WangTile { raw, user: None, conf: None, scene: None })
    }

    /// `WangTile::preload` (`wangtile.rs:340`): borrowed views of the merged splat texture and the static lists.
    pub fn preload(&mut self) -> Result<host::GswtPreload> {
        let mut p: host::GswtPreload = unsafe { std::mem::zeroed() };
        host_check(unsafe { host::gswt_wang_preload(self.raw, &mut p) })?;
        Ok(p)
    }

    /// `WangTile::configure` (`wangtile.rs:349`).
    pub fn configure(&mut self, user: &host::GswtUserData) -> Result<host::GswtConfigured> {
        let mut c: host::GswtConfigured = unsafe { std::mem::zeroed() };
        host_check(unsafe { host::gswt_wang_configure(self.raw, user, &mut c) })?;
        self.user = Some(*user);
        self.conf = Some(c);
        Ok(c)
    }

    /// `WangTile::check_update` (`wangtile.rs:692`).
    pub fn check_update(&self, cam_pos: [f32; 3]) -> bool {
        unsafe { host::gswt_wang_check_update(self.raw, cam_pos.as_ptr()) > 0 }
    }

    /// `WangTile::build_tiles` (`wangtile.rs:434`).
    pub fn build_tiles(&mut self, cam_pos: [f32; 3]) -> Result<host::GswtSceneData> {
        let mut s: host::GswtSceneData = unsafe { std::mem::zeroed() };
        host_check(unsafe { host::gswt_wang_build_tiles(self.raw, cam_pos.as_ptr(), &mut s) })?;
        self.scene = Some(s);
        Ok(s)
    }

    /// `WangTile::sort_tiles` (`wangtile.rs:476`).  The returned arrays are borrowed until the next mutating call.
    pub fn sort_tiles(&mut self, cam_pos: [f32; 3], view_proj: &[f32; 16]) -> Result<hip::GswtSortData> {
        let mut s: hip::GswtSortData = unsafe { std::mem::zeroed() };
        host_check(unsafe { host::gswt_wang_sort_tiles(self.raw, cam_pos.as_ptr(), view_proj.as_ptr(), &mut s) })?;
        Ok(s)
    }

    /// `SceneUniforms::from_data` (`renderer.rs:631-672`) for the current configuration and tile map.
    pub fn scene_uniforms(&self, splat_scale: f32, scene_scale: [f32; 3], height_map_scale_v: f32) -> Result<hip::GswtSceneUniforms> {
        let (user, conf, scene) = match (&self.user, &self.conf, &self.scene) {
            (Some(u), Some(c), Some(s)) => (u, c, s),
            _ => return Err(GswtError { code: hip::GSWT_ERR_STATE, message: "configure + build_tiles first".into() }),
        };
        let mut su: hip::GswtSceneUniforms = unsafe { std::mem::zeroed() };
        host_check(unsafe {
            host::gswt_scene_uniforms_from_data(user, conf, scene, splat_scale, scene_scale.as_ptr(), height_map_scale_v, &mut su)
        })?;
        Ok(su)
    }
}

impl Drop for WangTile {
    fn drop(&mut self) {
        unsafe { host::gswt_wang_destroy(self.raw) }
    }
}

/// `renderer::GSWTRenderer` on `libgswt_hip.so`: owns every device buffer; one caller thread at a time.
pub struct GSWTRenderer {
    ctx: *mut hip::GswtCtx,
}

impl GSWTRenderer {
    fn check(&self, rc: c_int) -> Result<()> {
        if rc == hip::GSWT_OK {
            return Ok(());
        }
        let msg = unsafe { CStr::from_ptr(hip::gswt_last_error(self.ctx)) };
        Err(GswtError { code: rc, message: msg.to_string_lossy().into_owned() })
    }

    /// `GSWTRenderer::new(&device, &queue, &config, wang.preload())` (`renderer.rs:31-349`): HIP device `device_id`
    /// instead of a wgpu device; uploads the Gaussian texture and every static base list.
    pub fn new(device_id: i32, preload: &host::GswtPreload) -> Result<Self> {
        let mut ctx: *mut hip::GswtCtx = ptr::null_mut();
        let rc = unsafe { hip::gswt_create(device_id, &mut ctx) };
        if rc != hip::GSWT_OK {
            return Err(GswtError { code: rc, message: "gswt_create failed (no HIP device?)".into() });
        }
        let r = GSWTRenderer { ctx };
        r.check(unsafe {
            hip::gswt_upload_scene(r.ctx, preload.tex_data, preload.n_splats, preload.lists, preload.n_lod, preload.n_tile, preload.n_view)
        })?;
        Ok(r)
    }

    /// `GSWTRenderer::configure` (`renderer.rs:351-405`): the height map (R32Float, linear, repeat) or none.
    pub fn configure(&mut self, conf: &host::GswtConfigured, surface_is_height_map: bool) -> Result<()> {
        let (p, w, h) = if surface_is_height_map {
            (conf.height_map, conf.height_map_wh[0] as c_int, conf.height_map_wh[1] as c_int)
        } else {
            (ptr::null(), 0, 0)
        };
        self.check(unsafe { hip::gswt_configure(self.ctx, p, w, h) })
    }

    /// Swap-in of a new `SortData` (`state.rs:361-376`): once per sort event, not per frame.
    pub fn set_sort_data(&mut self, sort: &hip::GswtSortData) -> Result<()> {
        let mut draws: Vec<hip::GswtDraw> = vec![unsafe { std::mem::zeroed() }; sort.n_tiles as usize];
        host_check(unsafe { host::gswt_renderer_build_draws(sort, draws.as_mut_ptr()) })?;
        self.check(unsafe {
            hip::gswt_set_draws(
                self.ctx, draws.as_ptr(), draws.len() as c_int, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id, sort.n_merged,
            )
        })
    }

    /// `GSWTRenderer::render` (`renderer.rs:407-592`).  `bg` = what the skybox / proxy passes left in the colour and depth
    /// attachments (host slices), `out` = `width * height * 4` floats (premultiplied RGBA, float framebuffer).
    pub fn render(
        &mut self,
        camera: &hip::GswtCameraUniforms,
        scene: &hip::GswtSceneUniforms,
        cfg: &hip::GswtRenderConfig,
        width: u32,
        height: u32,
        bg: Option<(&[f32], &[f32])>,
        out: &mut [f32],
    ) -> Result<()> {
        let n = (width as usize) * (height as usize);
        if out.len() < 4 * n {
            return Err(GswtError { code: hip::GSWT_ERR_BAD_ARG, message: "output slice too small".into() });
        }
        let (bg_rgba, bg_depth) = match bg {
            Some((c, d)) if c.len() >= 4 * n && d.len() >= n => (c.as_ptr(), d.as_ptr()),
            Some(_) => return Err(GswtError { code: hip::GSWT_ERR_BAD_ARG, message: "background slices too small".into() }),
            None => (ptr::null(), ptr::null()),
        };
        self.check(unsafe {
            hip::gswt_render(self.ctx, camera, scene, cfg, width as c_int, height as c_int, bg_rgba, bg_depth, 0, out.as_mut_ptr(), 0)
        })
    }

    /// Pipelined form (device pointers): up to `gswt_frame_slots()` frames overlap on the GPU.
    pub fn render_async(
        &mut self,
        camera: &hip::GswtCameraUniforms,
        scene: &hip::GswtSceneUniforms,
        cfg: &hip::GswtRenderConfig,
        width: u32,
        height: u32,
        bg_rgba_dev: *const f32,
        bg_depth_dev: *const f32,
        out_rgba_dev: *mut f32,
    ) -> Result<i32> {
        let mut ticket: c_int = -1;
        self.check(unsafe {
            hip::gswt_render_async(self.ctx, camera, scene, cfg, width as c_int, height as c_int, bg_rgba_dev, bg_depth_dev, out_rgba_dev, &mut ticket)
        })?;
        Ok(ticket)
    }

    /// Orders the ctx stream behind the frame; never releases an overflowed frame (see `gswt_hip.h`).
    pub fn render_fence(&mut self, ticket: i32) -> Result<()> {
        self.check(unsafe { hip::gswt_render_fence(self.ctx, ticket) })
    }

    pub fn render_wait(&mut self, ticket: i32) -> Result<hip::GswtTimings> {
        self.check(unsafe { hip::gswt_render_wait(self.ctx, ticket) })?;
        let mut t: hip::GswtTimings = unsafe { std::mem::zeroed() };
        self.check(unsafe { hip::gswt_last_timings(self.ctx, &mut t) })?;
        Ok(t)
    }
}

impl Drop for GSWTRenderer {
    fn drop(&mut self) {
        unsafe { hip::gswt_destroy(self.ctx) }
    }
}
