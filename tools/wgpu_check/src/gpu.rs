//! The reference's GPU seam, rebuilt with the reference's own calls (clientdesktop/src/graphics/mod.rs:213-275 device,
//! shader.rs:55-72 buffers, :301-344 pipeline and bind group, :371-379 the pass; main.rs:452 the dispatch): one device, the
//! buffers of bindings 1-3 and 5-7 filled from a scene dump, the rgba8unorm storage texture of binding 0, a dispatch of
//! width / 8 x height / 8 workgroups, and the texture (first pass) or the debug records (second pass) read back.
//! Written against wgpu 27.0.1 (the reference's Cargo.lock); this repository's environment has no cargo, so it has not been compiled
//! there — whatever drifts is in this file.

use crate::patch::{DEBUG_BINDING, DEBUG_ENTRY, DEBUG_RECORD_BYTES};
use crate::scene::Scene;
use wgpu::*;

pub struct Gpu {
    pub device: Device,
    pub queue: Queue,
    pub adapter_info: AdapterInfo,
}

impl Gpu {
    /// As `Gpu::new` (mod.rs:236-275) without a surface: default instance, default adapter, the adapter's own storage-binding and
    /// buffer-size limits.
    pub fn new() -> Gpu {
        pollster::block_on(async {
            let instance = Instance::new(&Default::default());
            let adapter = instance
                .request_adapter(&RequestAdapterOptions { power_preference: PowerPreference::default(), compatible_surface: None, force_fallback_adapter: false })
                .await
                .expect("no adapter");
            let limits = adapter.limits();
            let (device, queue) = adapter
                .request_device(&DeviceDescriptor {
                    required_limits: Limits {
                        max_storage_buffer_binding_size: limits.max_storage_buffer_binding_size,
                        max_buffer_size: limits.max_buffer_size,
                        ..Default::default()
                    },
                    ..Default::default()
                })
                .await
                .expect("no device");
            Gpu { device, queue, adapter_info: adapter.get_info() }
        })
    }

    fn wait(&self) {
        // wgpu 27: PollType::wait_indefinitely(); wgpu 25 / 26: PollType::Wait
        let _ = self.device.poll(PollType::wait_indefinitely());
    }

    fn buffer(&self, label: &str, usage: BufferUsages, bytes: &[u8], capacity: u64) -> Buffer {
        let b = self.device.create_buffer(&BufferDescriptor { label: Some(label), size: capacity.max(bytes.len() as u64).max(4), usage: usage | BufferUsages::COPY_DST, mapped_at_creation: false });
        if !bytes.is_empty() {
            self.queue.write_buffer(&b, 0, bytes);
        }
        b
    }

    fn read_back(&self, src: &Buffer, bytes: u64) -> Vec<u8> {
        let slice = src.slice(..bytes);
        slice.map_async(MapMode::Read, |r| r.expect("map failed"));
        self.wait();
        let data = slice.get_mapped_range().to_vec();
        src.unmap();
        data
    }
}

fn uniform() -> BindingType {
    BindingType::Buffer { ty: BufferBindingType::Uniform, has_dynamic_offset: false, min_binding_size: None }
}
fn storage(read_only: bool) -> BindingType {
    BindingType::Buffer { ty: BufferBindingType::Storage { read_only }, has_dynamic_offset: false, min_binding_size: None }
}
fn entry(binding: u32, ty: BindingType) -> BindGroupLayoutEntry {
    BindGroupLayoutEntry { binding, visibility: ShaderStages::COMPUTE, ty, count: None }
}

pub struct PassOutput {
    /// first pass: the result texture, tightly packed rgba8 rows
    pub rgba8: Vec<u8>,
    /// second pass: DEBUG_RECORD_BYTES per pixel (empty for the first pass)
    pub records: Vec<u8>,
}

/// One pass of `shader` over the scene: entry point `update` with the reference's bind group layout (`debug` false), or the
/// instrumented text's debug entry with one more storage buffer.
pub fn run_pass(gpu: &Gpu, scene: &Scene, shader: &str, debug: bool) -> PassOutput {
    let d = &gpu.device;
    let (w, h) = (scene.width, scene.height);
    let module = d.create_shader_module(ShaderModuleDescriptor { label: Some("ray_tracer.wgsl"), source: ShaderSource::Wgsl(shader.into()) });
    let tex = d.create_texture(&TextureDescriptor {
        label: Some("result"),
        size: Extent3d { width: w, height: h, depth_or_array_layers: 1 },
        mip_level_count: 1,
        sample_count: 1,
        dimension: TextureDimension::D2,
        format: TextureFormat::Rgba8Unorm,
        usage: TextureUsages::COPY_DST | TextureUsages::COPY_SRC | TextureUsages::STORAGE_BINDING | TextureUsages::TEXTURE_BINDING,
        view_formats: &[],
    });
    let view = tex.create_view(&TextureViewDescriptor::default());
    let cam = gpu.buffer("cam_data", BufferUsages::UNIFORM, &scene.cam_data, 0);
    let settings = gpu.buffer("settings", BufferUsages::UNIFORM, &scene.settings, 0);
    let world = gpu.buffer("world_data", BufferUsages::UNIFORM, &scene.world_data, 0);
    let mats = gpu.buffer("voxel_mats", BufferUsages::STORAGE, &scene.materials, 0);
    // the whole NodeBuffer (max_nodes / 2 words, zero beyond what was written), not only the words in use: what lies beyond the
    // pool's end is the backend's business, as in the reference
    let nodes = gpu.buffer("nodes", BufferUsages::STORAGE, &scene.node_words, scene.max_nodes as u64 * 2);
    let roots = gpu.buffer("chunk_roots", BufferUsages::STORAGE, &scene.chunk_roots, 0);
    let n_px = (w * h) as u64;
    let records = d.create_buffer(&BufferDescriptor { label: Some("debug records"), size: (n_px * DEBUG_RECORD_BYTES as u64).max(DEBUG_RECORD_BYTES as u64), usage: BufferUsages::STORAGE | BufferUsages::COPY_SRC, mapped_at_creation: false });

    let mut layout_entries = vec![
        entry(0, BindingType::StorageTexture { access: StorageTextureAccess::WriteOnly, format: TextureFormat::Rgba8Unorm, view_dimension: TextureViewDimension::D2 }),
        entry(1, uniform()),
        entry(2, uniform()),
        entry(3, storage(true)),
        entry(5, uniform()),
        entry(6, storage(true)),
        entry(7, storage(true)),
    ];
    let mut group_entries = vec![
        BindGroupEntry { binding: 0, resource: BindingResource::TextureView(&view) },
        BindGroupEntry { binding: 1, resource: cam.as_entire_binding() },
        BindGroupEntry { binding: 2, resource: settings.as_entire_binding() },
        BindGroupEntry { binding: 3, resource: mats.as_entire_binding() },
        BindGroupEntry { binding: 5, resource: world.as_entire_binding() },
        BindGroupEntry { binding: 6, resource: nodes.as_entire_binding() },
        BindGroupEntry { binding: 7, resource: roots.as_entire_binding() },
    ];
    if debug {
        layout_entries.push(entry(DEBUG_BINDING, storage(false)));
        group_entries.push(BindGroupEntry { binding: DEBUG_BINDING, resource: records.as_entire_binding() });
    }
    let bgl = d.create_bind_group_layout(&BindGroupLayoutDescriptor { label: Some("pixel shader"), entries: &layout_entries });
    let group = d.create_bind_group(&BindGroupDescriptor { label: Some("pixel shader"), layout: &bgl, entries: &group_entries });
    let layout = d.create_pipeline_layout(&PipelineLayoutDescriptor { label: None, bind_group_layouts: &[&bgl], push_constant_ranges: &[] });
    let pipeline = d.create_compute_pipeline(&ComputePipelineDescriptor {
        label: None,
        layout: Some(&layout),
        module: &module,
        entry_point: Some(if debug { DEBUG_ENTRY } else { "update" }),
        compilation_options: Default::default(),
        cache: None,
    });

    // the texture's rows, padded to the copy's row alignment
    let row = (w * 4).div_ceil(COPY_BYTES_PER_ROW_ALIGNMENT) * COPY_BYTES_PER_ROW_ALIGNMENT;
    let staging_bytes = if debug { n_px * DEBUG_RECORD_BYTES as u64 } else { row as u64 * h as u64 };
    let staging = d.create_buffer(&BufferDescriptor { label: Some("read-back"), size: staging_bytes.max(4), usage: BufferUsages::MAP_READ | BufferUsages::COPY_DST, mapped_at_creation: false });
    let mut enc = d.create_command_encoder(&CommandEncoderDescriptor { label: None });
    {
        let mut pass = enc.begin_compute_pass(&ComputePassDescriptor { label: Some("raytracer pass"), timestamp_writes: None });
        pass.set_pipeline(&pipeline);
        pass.set_bind_group(0, &group, &[]);
        pass.dispatch_workgroups(w / 8, h / 8, 1);   // main.rs:452
    }
    if debug {
        enc.copy_buffer_to_buffer(&records, 0, &staging, 0, staging_bytes);
    } else {
        enc.copy_texture_to_buffer(
            TexelCopyTextureInfo { texture: &tex, mip_level: 0, origin: Origin3d::ZERO, aspect: TextureAspect::All },
            TexelCopyBufferInfo { buffer: &staging, layout: TexelCopyBufferLayout { offset: 0, bytes_per_row: Some(row), rows_per_image: Some(h) } },
            Extent3d { width: w, height: h, depth_or_array_layers: 1 },
        );
    }
    gpu.queue.submit([enc.finish()]);
    let raw = gpu.read_back(&staging, staging_bytes);
    if debug {
        return PassOutput { rgba8: Vec::new(), records: raw };
    }
    let mut rgba8 = Vec::with_capacity((w * h * 4) as usize);
    for y in 0..h as usize {
        rgba8.extend_from_slice(&raw[y * row as usize..y * row as usize + (w * 4) as usize]);
    }
    PassOutput { rgba8, records: Vec::new() }
}
